export ENGINES=table SKIP_PER_EX=1 CUTS=default
for lib in "" build/libmfa_pf.so "" build/libmfa_pf.so; do
  echo "== lib=${lib:-default}"
  MFA_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/region_time.py 2>&1 | grep "^region"
  MFA_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/walk_check.py timing 2>&1 | grep -E "^mixed"
done
