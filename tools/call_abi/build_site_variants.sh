#!/bin/bash
# Second bisect step: WHICH call sites of the shared multiplier break when their callee has the product-scanning body.
# A second out-of-line instance fp_mul_call_ps (product scanning) is added next to fp_mul_call (operand scanning) and one
# group of call sites at a time is pointed at it.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/ark-blst_amd/csrc
OUT=$ROOT/ab_libs/call_abi
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DMI_TEST_HOOKS"
mkdir -p $OUT/isa
rm -f $OUT/lib_*.so
build() {   # name, file, sed expression
  local name=$1 file=$2 expr=$3; shift 3
  local w=$OUT/src_$name/ark-blst_amd/csrc
  rm -rf $OUT/src_$name; mkdir -p $w $OUT/src_$name/include
  cp $SRC/*.cuh $SRC/*.hpp $SRC/*.h $SRC/*.hip $w/; cp $ROOT/include/arkblst_amd.h $OUT/src_$name/include/
  # the second instance, right behind the first
  sed -i 's/^static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul_os(a, b); }/&\nstatic __device__ __noinline__ Fp fp_mul_call_ps(Fp a, Fp b) { return fp_mul(a, b); }/' $w/fp28.cuh
  sed -i 's/^FP_HD_NOINLINE Fp fp_mul_call(const Fp& a, const Fp& b) { return fp_mul_os(a, b); }/&\nFP_HD_NOINLINE Fp fp_mul_call_ps(const Fp\& a, const Fp\& b) { return fp_mul(a, b); }/' $w/fp28.cuh
  sed -i "$expr" $w/$file
  (cd $w && $HIPCC $FLAGS "$@" -c -o $OUT/pairing_api_$name.o pairing_api.hip &&
   $HIPCC $FLAGS "$@" --cuda-device-only -S -o $OUT/isa/$name.s pairing_api.hip 2>/dev/null)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $OUT/pairing_api_$name.o \
     $ROOT/ark-blst_amd/build/{api,msm_sort,msm_g1,msm_g2,points}.test.o
  rm -rf $OUT/src_$name $OUT/pairing_api_$name.o
  echo built $name
}
build site_mul_fp pairing.cuh '/static FP_HD E mul_fp/s/fp_mul_call/fp_mul_call_ps/g' &
build site_norm2  pairing.cuh '/static FP_HD E norm2/,/^    }/s/fp_mul_call/fp_mul_call_ps/g' &
build site_m1     ec.cuh      '/static FP_HD Fp m1/,/^    }/s/fp_mul_call/fp_mul_call_ps/g' &
build site_blst   fp28.cuh    '/^FP_HD Fp fp_from_blst\|^FP_HD void fp_to_blst\|^FP_HD bool fp_is_zero_any/s/fp_mul_call/fp_mul_call_ps/g' &
wait
ls -la $OUT
