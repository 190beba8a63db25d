#!/bin/bash
# Builds make_hdf5_fixtures.c against the libhdf5 of the build image (/opt/conda: HDF5 1.10.6) and writes tests/golden/hdf5/*.h5.
# The fixtures are committed; this only needs re-running to change them.  H5PREFIX overrides the install prefix.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
H5PREFIX="${H5PREFIX:-/opt/conda}"
mkdir -p "$here/hdf5"
tmp="$(mktemp -d)"
gcc -O1 -o "$tmp/mk" "$here/make_hdf5_fixtures.c" -I"$H5PREFIX/include" -L"$H5PREFIX/lib" -lhdf5 -Wl,-rpath,"$H5PREFIX/lib"
"$tmp/mk" "$here/hdf5"
rm -rf "$tmp"
ls -la "$here/hdf5"
