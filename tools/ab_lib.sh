#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_lib.sh <old.so> <python script> [args]   (the in-tree build is "new").
# The old build is selected through ECAMP_LIB; the product file is never touched.
OLD=$(readlink -f $1); shift
echo "== new"; python3 "$@"
echo "== old"; ECAMP_LIB=$OLD python3 "$@"
echo "== new again"; python3 "$@"
