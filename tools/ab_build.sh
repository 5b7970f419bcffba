#!/bin/bash
# Build a second libvct_hip (same ABI) with ONE csrc file taken from a git revision, for same-box A/B runs:
#   tools/ab_build.sh <git-rev> <file under csrc/> [more files]   ->  video-captioning-transformer_amd/libvct_hip_ab.so
# then:  VCT_LIB_PATH=$PWD/video-captioning-transformer_amd/libvct_hip_ab.so python bench.py ...
set -e
rev=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
tmp=/tmp/vct_ab_$$
rm -rf $tmp; mkdir -p $tmp/pkg/csrc $tmp/include
cp $root/video-captioning-transformer_amd/csrc/* $tmp/pkg/csrc/
cp $root/include/vct_hip.h $tmp/include/
for f in "$@"; do git -C $root show $rev:video-captioning-transformer_amd/csrc/$f > $tmp/pkg/csrc/$f; done
cd $tmp/pkg/csrc
objs=""
for s in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -c $s -o ${s%.hip}.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC *.o -ldl -o $root/video-captioning-transformer_amd/libvct_hip_ab.so
echo built $root/video-captioning-transformer_amd/libvct_hip_ab.so
