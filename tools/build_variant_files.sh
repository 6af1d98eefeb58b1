#!/bin/bash
# tools/build_variant_files.sh name "flags" base(main|prof) file1 [file2 ...]: like build_variants.sh's variant form, several sources rebuilt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SNAP=${SNAP:-/tmp/erl_build_snap}
name=$1; flags=$2; base=$3; shift 3
cd $SNAP/pkg/elegantrl_amd/csrc
bdir=build; bflags=""; [ "$base" = prof ] && { bdir=build_prof; bflags="-DERL_PROFILE"; }
rm -rf build_$name; cp -rp $bdir build_$name
for f in "$@"; do rm -f build_$name/$f.o; done
make -j8 OBJDIR=build_$name EXTRA="$bflags $flags" OUT=$ROOT/elegantrl_amd/lib/liberl_hip_$name.so > $SNAP/$name.log 2>&1 && echo "$name ok" || { echo "$name FAILED"; grep -m5 -B2 -A6 "error" $SNAP/$name.log; exit 1; }
