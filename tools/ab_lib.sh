#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_lib.sh <old.so> <python script> [args]   (the in-tree build is "new").
# The old build is selected through ECAMP_LIB; the product file is never touched.  A build of another ABI version is refused here
# (and again by ecamp_amd/_lib.py): ctypes would hand it this header's argument lists.
R=${GRAFT_REPO_ROOT:-/root/repo}
OLD=$(readlink -f $1); shift
WANT=$(sed -n 's/^#define ECAMP_ABI_VERSION \([0-9]*\).*/\1/p' $R/include/ecamp_hip.h)
GOT=$(python3 -c "import ctypes,sys; print(ctypes.CDLL(sys.argv[1]).ecamp_abi_version())" $OLD)
if [ "$GOT" != "$WANT" ]; then echo "ab_lib: $OLD has ABI version $GOT, the header declares $WANT -- refusing the A/B"; exit 2; fi
echo "== new"; python3 "$@"
echo "== old"; ECAMP_LIB=$OLD python3 "$@"
echo "== new again"; python3 "$@"
