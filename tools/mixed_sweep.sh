export ENGINES=table CUTS="0.3,0.6,0.8,0.9" SKIP_PER_EX=1
timeout -k 10 300 python tools/walk_check.py timing 2>&1 | grep -E "^mixed|^ex"
timeout -k 10 300 python tools/walk_bench.py table rev8 nonper 2>&1 | grep -v amdgpu
MFA_WALK_C=12 timeout -k 10 300 python tools/walk_bench.py table rev8 2>&1 | grep -v amdgpu
MFA_WALK_C=5 timeout -k 10 300 python tools/walk_bench.py table rev8 2>&1 | grep -v amdgpu
