# builds the library of another commit as tools/exp/libyolat_hip_prev.so (the "prev" side of tools/exp/r06_ab2.sh / r06_kstats_ab.sh)
# usage: bash tools/exp/r06_build_prev.sh <commit>      (build container: hipcc cross-compiles gfx950)
set -e
C=${1:?commit}
W=$(mktemp -d)
git worktree add --detach "$W" "$C" > /dev/null
make -C "$W/yolat_vectorgraphicsrecognition_amd/csrc" > /dev/null
cp "$W/yolat_vectorgraphicsrecognition_amd/libyolat_hip.so" tools/exp/libyolat_hip_prev.so
git worktree remove --force "$W"
ls -la tools/exp/libyolat_hip_prev.so
