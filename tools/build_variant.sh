#!/bin/bash
# Builds a second libisegmi.so with extra compiler flags into instancesegmentation-jittor_amd/lib_<name>/ (for tools/ab_lib.sh, ISEGMI_LIB=...).
#   tools/build_variant.sh <name> <flags...>
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/instancesegmentation-jittor_amd
tmp=$(mktemp -d)
cp -r $pkg/csrc $pkg/Makefile $tmp/
mkdir -p $tmp/../include && true
(cd $tmp && sed -i "s|../include/isegmi.h|$root/include/isegmi.h|g" Makefile && sed -i "s|\"../../include/isegmi.h\"|\"$root/include/isegmi.h\"|" csrc/*.hip csrc/*.cpp csrc/*.h &&
 make -j8 -s FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function $*")
mkdir -p $pkg/lib_$name && cp $tmp/lib/libisegmi.so $pkg/lib_$name/
rm -rf $tmp
echo "built $pkg/lib_$name/libisegmi.so with $*"
