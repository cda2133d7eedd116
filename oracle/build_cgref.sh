#!/bin/bash
# oracle/build_cgref.sh -- oracle/_ref/libcgklt_ref.so: the reference's KLT fragment programs
# (src/tracking/CGKLT/Shaders/*.cg) compiled where they lie, run by ref_shim/cg/cgklt_driver.cpp.
#
# TEST INFRASTRUCTURE.  Each .cg file is piped through sed into clang++ with ref_shim/cg/cg_shim.h force-included; no copy
# of a shader is written.  The rewrites (all syntactic, listed in cg_shim.h's header too):
#   ` : TEXUNITn / TEXCOORDn / COLOR`  binding semantics, stripped
#   `out T x`                          -> `T& x`
#   `(0).xxx`                          -> `float3(0)`   (swizzle of a scalar, Cg only; klt_tracker.cg:75-76)
#   `0.00001`                          -> `0.00001f`    (Cg's unsuffixed literals are binary32; klt_tracker.cg:113, _with_gain.cg:142)
# and the program is wrapped in a namespace named after its macro variant.  Macros come from where the host passes them:
# PRESMOOTHING (v3d_gpupyramid.cpp:340, = 1 from v3d_gpuklt.cpp:600-601), HALF_WIDTH / N_LEVELS / LEVEL_SKIP
# (v3d_gpuklt.cpp:108-115, 212), MIN_DIST (:380), PYR_LEVELS (:391).
set -e
cd "$(dirname "$0")"
REFERENCE=${REFERENCE:-/root/reference}
SH=$REFERENCE/src/tracking/CGKLT/Shaders
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang++}
[ -f "$SH/klt_tracker_with_gain.cg" ] || { echo "build_cgref: no reference shaders, skipped"; exit 0; }
[ -x "$CLANG" ] || { echo "build_cgref: no clang++ at $CLANG, skipped"; exit 0; }
OUT=_ref/cg
mkdir -p $OUT
FLAGS="-x c++ -std=c++14 -O2 -fPIC -ffp-contract=off -fno-fast-math -w -include ref_shim/cg/cg_shim.h"

GAIN="1 2 3 4 5 6 7"
NOGAIN=""
for L in 2 3 4 5 6; do for S in $(seq 1 $((L-1))); do for hw in 2 3 5; do NOGAIN="$NOGAIN $L,$S,$hw"; done; done; done
NONMAX="1 2 3 4 5 6 7 8 9 10 11 12"
TRAVERSE="3 4 5 6 7 8 9 10 11 12"

# one line per object: namespace, shader, macros
JOBS=$OUT/jobs.txt
{
  echo "cg_pyr_pass1v pyramid_with_derivative_pass1v -DPRESMOOTHING=1"
  echo "cg_pyr_pass1h pyramid_with_derivative_pass1h -DPRESMOOTHING=1"
  echo "cg_pyr_pass2 pyramid_with_derivative_pass2"
  echo "cg_det_pass1 klt_detector_pass1"
  echo "cg_det_pass2 klt_detector_pass2"
  echo "cg_det_discriminator klt_detector_discriminator"
  echo "cg_det_build_histpyr klt_detector_build_histpyr"
  for hw in $GAIN; do echo "cg_gain_hw$hw klt_tracker_with_gain -DHALF_WIDTH=$hw"; done
  for v in $NOGAIN; do IFS=, read L S hw <<< "$v"; echo "cg_nogain_L${L}_S${S}_hw$hw klt_tracker -DN_LEVELS=$L -DLEVEL_SKIP=$S -DHALF_WIDTH=$hw"; done
  for d in $NONMAX; do echo "cg_nonmax_d$d klt_detector_nonmax -DMIN_DIST=$d"; done
  for n in $TRAVERSE; do echo "cg_traverse_n$n klt_detector_traverse_histpyr -DPYR_LEVELS=$n"; done
} > $JOBS

one() {
  ns=$1; f=$2; shift 2
  { echo "namespace $ns {"
    sed -e 's/ *: *\(TEXUNIT[0-9]\|TEXCOORD[0-9]\|COLOR\)//' -e 's/\bout \(float[234]\?\) /\1\& /g' \
        -e 's/(0)\.xxx/float3(0)/g' -e 's/0\.00001\b/0.00001f/' "$SH/$f.cg"
    echo "}"; } | $CLANG $FLAGS "$@" -c - -o $OUT/$ns.o
}
export -f one; export SH CLANG FLAGS OUT
xargs -P "${CGREF_JOBS:-8}" -L 1 bash -c 'one "$@"' _ < $JOBS

xl() { for v in $1; do printf 'X(%s) ' "$v"; done; }
$CLANG -std=c++14 -O2 -fPIC -ffp-contract=off -fno-fast-math -w -Iref_shim/cg \
  "-DCG_GAIN_LIST=$(xl "$GAIN")" "-DCG_NOGAIN_LIST=$(xl "$NOGAIN")" "-DCG_NONMAX_LIST=$(xl "$NONMAX")" \
  "-DCG_TRAVERSE_LIST=$(xl "$TRAVERSE")" -shared ref_shim/cg/cgklt_driver.cpp $(awk '{print "'$OUT'/" $1 ".o"}' $JOBS) \
  -o _ref/libcgklt_ref.so
rm -f $JOBS
echo "built _ref/libcgklt_ref.so"
