#!/usr/bin/env bash
# build_ref.sh -- compile the reference's own CPU golden loops into oracle/_ref/.
#
# TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build container);
# the GPU box receives the built files in oracle/_ref/ and never the reference itself.
#
# The statements are cut from the reference sources where they lie, by line range, into a
# scratch directory that is removed afterwards; only machine code is kept:
#   oracle/_ref/libfdtd_ref_strict.so   -O2 -ffp-contract=off   (IEEE meaning of the source; parity pin)
#   oracle/_ref/ref_bench               -std=c++11 -Ofast -fopenmp  (CMakeLists.txt:10,17 host flags; CPU baseline)
set -euo pipefail
REF="${FDTD_REFERENCE_DIR:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then
    echo "build_ref.sh: $REF not present; keeping prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$OUT"
TMP="$(mktemp -d /tmp/fdtd_ref_frag.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT

cut_lines() {  # file first last out [must-start-with] [must-end-with]
    sed -n "${2},${3}p" "$REF/$1" > "$TMP/$4"
    if [ -n "${5:-}" ] && ! head -n1 "$TMP/$4" | grep -qF -- "$5"; then
        echo "build_ref.sh: $1:$2 does not start with '$5' -- reference layout changed" >&2; exit 1
    fi
    if [ -n "${6:-}" ] && ! tail -n1 "$TMP/$4" | grep -qF -- "$6"; then
        echo "build_ref.sh: $1:$3 does not end with '$6' -- reference layout changed" >&2; exit 1
    fi
}

cut_lines fdtd1.cu 12 12 frag_get_macro.inc "#define GET("
cut_lines fdtd1.cu 21 31 frag_init_mt.inc "#pragma omp parallel for" "}"
cut_lines main.cu 24 36 frag_init_mt3.inc "#pragma omp parallel for" "}"
for n in 1 2 3 4 5 6 7 8; do
    cut_lines "fdtd$n.cu" 42 $((46 + 3 * n)) "frag_star$n.inc" "#pragma omp parallel for" "));"
done
cut_lines fdtdspc.cu 42 52 frag_spc13.inc "#pragma omp parallel for" "));"
cut_lines jacobi.cu 15 21 frag_jacobi_consts.inc "const float cc" "const float cb"
cut_lines jacobi.cu 412 417 frag_jacobi_init.inc "for (int64_t k" "}"
cut_lines jacobi.cu 475 485 frag_jacobi_step.inc "for (int64_t k" "+cc*host_B"
cut_lines main.cu 54 84 frag_main_step.inc "#pragma omp parallel for" "sum[GET("

g++ -std=c++11 -O2 -ffp-contract=off -fopenmp -fPIC -shared -I"$TMP" \
    "$HERE/ref_harness.cpp" -o "$OUT/libfdtd_ref_strict.so"
g++ -std=c++11 -Ofast -fopenmp -DREF_BENCH_MAIN -I"$TMP" \
    "$HERE/ref_harness.cpp" -o "$OUT/ref_bench"
echo "build_ref.sh: built $OUT/libfdtd_ref_strict.so and $OUT/ref_bench"
