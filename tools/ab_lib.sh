#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_lib.sh <old.so> <python script> [args]   (the in-tree build is "new")
R=${GRAFT_REPO_ROOT:-/root/repo}
OLD=$1; shift
echo "== new"; python3 "$@"
cp $R/ecamp_amd/libecamp_hip.so /tmp/lib_new_keep.so; cp $OLD $R/ecamp_amd/libecamp_hip.so
echo "== old"; python3 "$@"
cp /tmp/lib_new_keep.so $R/ecamp_amd/libecamp_hip.so
echo "== new again"; python3 "$@"
