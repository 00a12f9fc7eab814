# same-box A/B of two builds of the library: the in-tree one and a previous one under cfd-proxy_amd/lib_prev (CFDP_LIBDIR),
# alternating processes.  usage: bash tools/ab_libs.sh [sizes...]
for rep in 1 2 3; do
  echo "== previous build"; CFDP_LIBDIR=$PWD/cfd-proxy_amd/lib_prev FORMS=${PREV_FORMS:-1} python tools/pass_time.py "$@" 2>&1 | grep "^n "
  echo "== this build";     FORMS=${FORMS:-2} python tools/pass_time.py "$@" 2>&1 | grep "^n "
done
