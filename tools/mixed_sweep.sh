export ENGINES=table SKIP_PER_EX=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/walk_check.py timing 2>&1 | grep -E "^mixed"; }
export CUTS="default|0.3,0.6,0.8,0.9"
run A=new-default
run MFA_REGION_BLOCK=256
run A=new-default-again
