#!/bin/bash
# Variant test libraries for the round-3 miscompare: product-scanning body behind fp28::fp_mul_call(Fp, Fp) made the test-only
# single-lane Miller kernel (k_miller_loop, pairing_api translation unit) disagree with the production path on the GPU.
# Only pairing_api.test.o is rebuilt per variant (fp_mul_call is static per translation unit); the other objects are the shipped ones.
#   usage: tools/call_abi/build_variants.sh   -> ab_libs/call_abi/lib_<variant>.so + <variant>.s (device ISA)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/ark-blst_amd/csrc
OUT=$ROOT/ab_libs/call_abi
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DMI_TEST_HOOKS"
mkdir -p $OUT
make -s -j8 -C $SRC
build() {   # name, sed expression on fp28.cuh, extra flags
  local name=$1 expr=$2; shift 2
  local w=$OUT/src_$name/ark-blst_amd/csrc
  rm -rf $OUT/src_$name; mkdir -p $w $OUT/src_$name/include
  cp $SRC/*.cuh $SRC/*.hpp $SRC/*.h $SRC/*.hip $w/; cp $ROOT/include/arkblst_amd.h $OUT/src_$name/include/
  sed -i "$expr" $w/fp28.cuh
  (cd $w && $HIPCC $FLAGS "$@" -c -o $OUT/pairing_api_$name.o pairing_api.hip &&
   $HIPCC $FLAGS "$@" --cuda-device-only -S -o $OUT/$name.s pairing_api.hip 2>/dev/null)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $OUT/pairing_api_$name.o \
     $ROOT/ark-blst_amd/build/{api,msm_sort,msm_g1,msm_g2,points}.test.o
  rm -rf $OUT/src_$name $OUT/pairing_api_$name.o
  echo built $name
}
PS='s/static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul_os(a, b); }/static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul(a, b); }/'
PSREF='s/static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul_os(a, b); }/static __device__ __noinline__ Fp fp_mul_call(const Fp\& a, const Fp\& b) { return fp_mul(a, b); }/'
build os        's/^$//' &
build ps        "$PS" &
build ps_noalias "$PS" -fno-strict-aliasing &
build ps_noipra "$PS" -mllvm -enable-ipra=false &
wait
build ps_wait0  "$PS" -mllvm -amdgpu-waitcnt-forcezero &
build ps_ref    "$PSREF" &
build ps_O1     "$PS" -O1 &
build os_noipra 's/^$//' -mllvm -enable-ipra=false &
wait
ls -la $OUT
