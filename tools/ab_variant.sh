# same-box A/B of the in-tree library against a variant build under cfd-proxy_amd/$1 (CFDP_LIBDIR), alternating processes
V=$1; shift
for rep in 1 2 3; do
  echo "== in-tree";  FORMS=2 python tools/pass_time.py "$@" 2>&1 | grep "^n "
  echo "== $V";       CFDP_LIBDIR=$PWD/cfd-proxy_amd/$V FORMS=2 python tools/pass_time.py "$@" 2>&1 | grep "^n "
done
