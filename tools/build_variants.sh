#!/bin/bash
# Builds liberl_hip variants from a SNAPSHOT of the sources (so that edits made while it runs do not end up half-compiled):
#   tools/build_variants.sh main prof [name:"extra flags":file.o-to-rebuild-from-base ...]
# main -> elegantrl_amd/lib/liberl_hip.so, prof -> liberl_hip_prof.so (-DERL_PROFILE); a variant "x2:-DERL_K6_EXP=2:main:ppo_step_s3_pre.hip"
# copies the objects of `main` (or `prof`), rebuilds the named source with the extra flags and links liberl_hip_x2.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SNAP=${SNAP:-/tmp/erl_build_snap}
mkdir -p $SNAP/pkg/elegantrl_amd/csrc $SNAP/pkg/elegantrl_amd/lib $SNAP/pkg/include
# copy a source only when its content differs (mtimes of unchanged files survive: make stays incremental)
python3 - "$ROOT" "$SNAP/pkg" <<'PY'
import filecmp, os, shutil, sys
root, snap = sys.argv[1], sys.argv[2]
for sub in ("elegantrl_amd/csrc", "include"):
    src, dst = os.path.join(root, sub), os.path.join(snap, sub)
    names = [n for n in os.listdir(src) if os.path.isfile(os.path.join(src, n))]
    for n in names:
        a, b = os.path.join(src, n), os.path.join(dst, n)
        if not os.path.exists(b) or not filecmp.cmp(a, b, shallow=False):
            shutil.copy2(a, b)
    for n in os.listdir(dst):
        if os.path.isfile(os.path.join(dst, n)) and n not in names:
            os.remove(os.path.join(dst, n))
PY
cd $SNAP/pkg/elegantrl_amd/csrc
for v in "$@"; do
  case $v in
    main) make -j8 OBJDIR=build OUT=$ROOT/elegantrl_amd/lib/liberl_hip.so > $SNAP/main.log 2>&1 && echo "main ok" || { echo "main FAILED"; grep -m5 -B2 -A6 "error" $SNAP/main.log; exit 1; } ;;
    prof) make -j8 OBJDIR=build_prof EXTRA=-DERL_PROFILE OUT=$ROOT/elegantrl_amd/lib/liberl_hip_prof.so > $SNAP/prof.log 2>&1 && echo "prof ok" || { echo "prof FAILED"; grep -m5 -B2 -A6 "error" $SNAP/prof.log; exit 1; } ;;
    *)
      name=${v%%:*}; rest=${v#*:}; flags=${rest%%:*}; rest=${rest#*:}; base=${rest%%:*}; file=${rest#*:}
      bdir=build; bflags=""; [ "$base" = prof ] && { bdir=build_prof; bflags="-DERL_PROFILE"; }
      rm -rf build_$name; cp -rp $bdir build_$name; rm -f build_$name/$file.o
      make -j8 OBJDIR=build_$name EXTRA="$bflags $flags" OUT=$ROOT/elegantrl_amd/lib/liberl_hip_$name.so > $SNAP/$name.log 2>&1 && echo "$name ok" || { echo "$name FAILED"; grep -m5 -B2 -A6 "error" $SNAP/$name.log; exit 1; } ;;
  esac
done
